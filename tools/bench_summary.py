"""Print the parts of a bench.py JSON line that a round's notes quote (tools/gpu_run.sh plan `bench`)."""
import json
import sys

for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        if len(sys.argv) > 2:
            json.dump(d, open(sys.argv[2], "w"), indent=1)
        print("bench", d["value"], d["ms_per_step"], "first chunk", d.get("first_chunk_ms_p50"), d.get("stages"))
        print("   self_check", {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if not isinstance(vv, (list, dict))}) for k, v in d.get("self_check", {}).items()})
        for k in ("batched_decode", "batched_decode_16", "batched_decode_32", "streaming_clients", "mixed64", "cosyvoice3", "cosyvoice300m"):
            print("  ", k, json.dumps(d.get(k))[:900])
        r = d["roofline"]
        print("   roofline", {k: r.get(k) for k in ("achieved", "frac", "frac_kernel_trace", "avg_launch_us", "traffic", "decode_stage", "decode_step_us_from_chains")})
        if "roofline_mfma" in d:
            print("   roofline_mfma", d["roofline_mfma"])
        if "cpu_baseline" in d:
            print("   cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind", "stage_seconds")})
