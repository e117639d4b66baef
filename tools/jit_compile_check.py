"""Compile (hipcc, gfx950, no GPU needed) the run-time sources the JIT path would hand to hiprtc for the join kernels:
build (5), key range (14), chained probe (6), unique probe (7, generic and the one-narrow-key specialisation)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import arrow_ballista_amd as g
from arrow_ballista_amd.expr import col, lit, binary, Operator as Op

fields = [{"name": "k", "type": "Int64", "nullable": False}, {"name": "d", "type": "Date32", "nullable": False}]
pred = binary(col("d", fields), Op.Gt, lit(9204, "Date32"))
build = {"op": "join_build", "input": {"fields": fields}, "on": [col("k", fields)], "predicate": pred}
probe = {"op": "join_probe", "input": {"fields": fields}, "on": [col("k", fields)], "predicate": pred, "join_type": "Inner"}
jobs = [(build, 5, ""), (build, 14, ""), (probe, 15, ""), (probe, 6, ""), (probe, 7, ""), (probe, 7, "#define GPUQ_JIT_PROBE1 1\nconstexpr int JIT_KEY_REG0 = %d;\n")]
with tempfile.TemporaryDirectory() as d:
    for desc, kid, spec in jobs:
        src = g.compile_jit_source(desc, kid)
        if spec:
            import json, re
            chk = g.binding.compile_check(desc)
            m = re.search(r"key_regs?\D+(\d+)", json.dumps(chk))
            kr = 0
            # the key register is the one the program leaves the key in: take it from the description when present
            for k_ in ("key_reg", "key_regs"):
                if k_ in chk:
                    kr = chk[k_][0] if isinstance(chk[k_], list) else chk[k_]
            marker = '}\n#include "kernels_hash.hip"'
            assert marker in src
            src = src.replace(marker, (spec % kr) + marker)
        f = os.path.join(d, "k%d.hip" % kid)
        open(f, "w").write(src)
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", "-x", "hip", f, "-Rpass-analysis=kernel-resource-usage",
                            "-I", os.path.join(ROOT, "arrow-ballista_amd", "csrc"), "-o", os.path.join(d, "o.o")], capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-3000:]); sys.exit(1)
        res = [l.split("remark: ")[-1].strip() for l in r.stderr.splitlines() if any(t in l for t in ("Function Name", "VGPRs:", "ScratchSize", "SGPRs:", "Occupancy"))]
        print("kernel %d%s OK: %s" % (kid, " (PROBE1)" if spec else "", "; ".join(res[-5:])))
