"""Compile (hipcc, gfx950, no GPU needed) the run-time sources the JIT path would hand to hiprtc, one per kernel id:
filter (1), project (2), tiny aggregate (3), hash aggregate (4, 11-13), join build (5), key range (14), radix-join pack (15),
chained probe (6), unique probe (7, generic and the one-narrow-key specialisation), sort min/max (8), sort pack (9),
partition ids (10).  A source that does not compile makes the operator fall back to its AOT kernel -- silently slower."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import arrow_ballista_amd as g
from arrow_ballista_amd.expr import col, lit, binary, Operator as Op

# hiprtc has no libc / libstdc++ headers: a system #include that is live under GPUQ_JIT compiles here (hipcc) and fails on the GPU box,
# where the operator then silently runs its AOT kernel.  Walk the files a run-time translation unit includes and refuse any such line.
def live_includes_under_jit(path):
    bad, stack = [], []          # stack of booleans: is this conditional branch live when GPUQ_JIT is defined?
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith("#ifndef GPUQ_JIT"): stack.append(False)
        elif t.startswith("#ifdef GPUQ_JIT") or t.startswith("#if defined(GPUQ_JIT)"): stack.append(True)
        elif t.startswith("#if"): stack.append(all(stack) if stack else True)
        elif t.startswith("#else") and stack: stack[-1] = not stack[-1] if t == "#else" else stack[-1]
        elif t.startswith("#elif") and stack: stack[-1] = True if "GPUQ_JIT_KERNEL" in t else stack[-1]
        elif t.startswith("#endif") and stack: stack.pop()
        elif t.startswith("#include <") and all(stack) and not any(h in t for h in ("hip/hip_runtime.h", "stdint.h")): bad.append("%s:%d: %s" % (os.path.basename(path), ln, t))
    return bad
CSRC = os.path.join(ROOT, "arrow-ballista_amd", "csrc")
bad = sum((live_includes_under_jit(os.path.join(CSRC, f)) for f in ("kernels_scan.hip", "kernels_hash.hip", "kernels_sort.hip", "gpuq_dev.h", "gpuq_kernels.h")), [])
if bad:
    print("system headers live under GPUQ_JIT (hiprtc cannot find them):\n  " + "\n  ".join(bad)); sys.exit(1)

fields = [{"name": "k", "type": "Int64", "nullable": False}, {"name": "d", "type": "Date32", "nullable": False}]
pred = binary(col("d", fields), Op.Gt, lit(9204, "Date32"))
build = {"op": "join_build", "input": {"fields": fields}, "on": [col("k", fields)], "predicate": pred}
probe = {"op": "join_probe", "input": {"fields": fields}, "on": [col("k", fields)], "predicate": pred, "join_type": "Inner"}

proj = {"op": "project", "input": {"fields": fields}, "exprs": [{"expr": binary(col("k", fields), Op.Plus, lit(1, "Int64")), "name": "k1"}]}
filt = {"op": "filter", "input": {"fields": fields}, "predicate": pred}
aggr = {"op": "aggregate", "input": {"fields": fields}, "mode": "Single", "group_expr": [{"expr": col("d", fields), "name": "d"}],
        "aggr_expr": [{"fn": "SUM", "expr": col("k", fields), "name": "s"}, {"fn": "COUNT", "expr": col("k", fields), "name": "c"}]}
sort = {"op": "sort", "input": {"fields": fields}, "expr": [{"expr": col("k", fields), "asc": False, "nulls_first": False}, {"expr": col("d", fields), "asc": True, "nulls_first": False}]}
part = {"op": "partition", "input": {"fields": fields}, "hash_expr": [col("k", fields)], "partition_count": 16}
fields3 = fields + [{"name": "k2", "type": "Int64", "nullable": False}]
build_semi = {"op": "join_build", "input": {"fields": fields3}, "on": [col("k", fields3)], "semi_on": [col("k2", fields3)], "predicate": binary(col("d", fields3), Op.Gt, lit(9204, "Date32"))}
jobs = [(build_semi, 5, "#define GPUQ_JIT_SEMI 1\n"), (build_semi, 5, "#define GPUQ_JIT_SEMI 2\nconstexpr int JIT_KEY_REG0 = 1;\nconstexpr int JIT_KEY2_REG = 0;\n"), (filt, 1, ""), (proj, 2, ""), (aggr, 3, ""), (aggr, 4, ""), (aggr, 11, ""), (aggr, 12, ""), (aggr, 13, ""), (sort, 8, ""), (sort, 9, ""), (part, 10, ""), (build, 5, ""), (build, 14, ""), (probe, 15, ""), (probe, 6, ""), (probe, 7, ""), (probe, 7, "#define GPUQ_JIT_PROBE1 1\nconstexpr int JIT_KEY_REG0 = %d;\n")]
with tempfile.TemporaryDirectory() as d:
    for desc, kid, spec in jobs:
        src = g.compile_jit_source(desc, kid)
        if spec and "GPUQ_JIT_SEMI" in spec:
            marker = '}\n#include "kernels_hash.hip"'
            assert marker in src
            src = src.replace(marker, spec + marker)
        elif spec:
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
        print("kernel %d%s OK: %s" % (kid, " (SEMI)" if "SEMI" in spec else " (PROBE1)" if spec else "", "; ".join(res[-5:])))
