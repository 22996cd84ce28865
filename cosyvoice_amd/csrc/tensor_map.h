// Named device tensors registered through cv_*_set_tensor (weights packed by cosyvoice_amd/weights.py).
#pragma once
#include <map>
#include <string>
#include "api_common.h"

namespace cv {

struct Tensor { const void* p = nullptr; int dtype = CV_F32; long long numel = 0; };

struct TensorMap {
    std::map<std::string, Tensor> t;
    void set(const char* name, const void* p, int dtype, long long numel) {
        CV_CHECK(name && p && numel > 0, "set_tensor: bad arguments");
        CV_CHECK(aligned16(p), std::string("set_tensor: ") + name + " must be 16B aligned");
        t[name] = Tensor{p, dtype, numel};
    }
    const Tensor& get(const std::string& name, int dtype, long long numel) const {
        auto it = t.find(name);
        if (it == t.end()) throw Error("missing tensor '" + name + "'");
        if (it->second.dtype != dtype) throw Error("tensor '" + name + "' has the wrong dtype");
        if (numel >= 0 && it->second.numel != numel)
            throw Error("tensor '" + name + "' has " + std::to_string(it->second.numel) + " elements, expected " + std::to_string(numel));
        return it->second;
    }
    const float* f32(const std::string& name, long long numel) const { return reinterpret_cast<const float*>(get(name, CV_F32, numel).p); }
    const unsigned short* bf16(const std::string& name, long long numel) const { return reinterpret_cast<const unsigned short*>(get(name, CV_BF16, numel).p); }
    bool has(const std::string& name) const { return t.count(name) != 0; }
};

static inline int round_up32(int k) { return (k + 31) / 32 * 32; }

}  // namespace cv
