// Dev tool: does the row pitch of a tile read matter (L2 channel camping)?  344 workgroups of 256 threads each read NIT tiles of
// 32 rows x 1 KB (one wave-wide 16 B/lane load per row and iteration) from a matrix with row pitch P bytes; all workgroups walk
// the k direction in the same order ("lockstep") or start at a per-workgroup offset ("staggered").  Prints aggregate GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void read_tiles(const float* A, long long pitch_f, int rows_total, int nit, int stagger, float* sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = (blockIdx.x % (rows_total / 32)) * 32;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int shift = stagger ? (blockIdx.x * 5) % nit : 0;
    for (int it = 0; it < nit; ++it) {
        const int kc = (it + shift) % nit;
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {                 // 4 waves x 8 rows = 32 rows, 1 KB each
            const int row = m0 + wave * 8 + i;
            v[i] = *reinterpret_cast<const float4*>(A + (long long)row * pitch_f + kc * 256 + lane * 4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
    if (acc.x == 12345.f) sink[0] = acc.y + acc.z + acc.w;
}

int main() {
    const int rows = 1344, nblk = 344 * 1;
    float* sink; (void)hipMalloc(&sink, 16);
    for (int K : {1024, 256, 1536}) {
        const int nit = K / 256 > 0 ? K / 256 : 1;
        for (int pad : {0, 64, 16}) {
            const long long pitch_f = K + pad;
            float* A; (void)hipMalloc(&A, (size_t)rows * pitch_f * 4 + 4096); (void)hipMemset(A, 0, (size_t)rows * pitch_f * 4 + 4096);
            for (int stagger = 0; stagger < 2; ++stagger) {
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                const int reps = 50, loops = 8;            // every workgroup re-walks its band `loops` times (L2-resident after the first)
                for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(read_tiles, dim3(nblk), dim3(256), 0, 0, A, pitch_f, rows, nit * loops, stagger, sink);
                (void)hipEventRecord(e0, 0);
                for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(read_tiles, dim3(nblk), dim3(256), 0, 0, A, pitch_f, rows, nit * loops, stagger, sink);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)nblk * nit * loops * 32 * 1024 * reps;
                printf("K=%4d pitch=%5lld B  %-9s  %7.1f GB/s  (%.1f us per launch)\n", K, pitch_f * 4, stagger ? "staggered" : "lockstep", bytes / (ms * 1e-3) / 1e9, ms * 1e3 / reps);
            }
            (void)hipFree(A);
        }
    }
    return 0;
}
