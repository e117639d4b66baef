"""Dump the run-time (hiprtc) source of the q1 partial-aggregate kernel and its gfx950 ISA statistics (no GPU needed)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import arrow_ballista_amd as g, tpch_util as T
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/q1jit"
os.makedirs(out, exist_ok=True)
cols = [g.DeviceColumn(n, t, None, 0, nullable=False) for n, t in [("l_quantity", T.D152), ("l_extendedprice", T.D152), ("l_discount", T.D152), ("l_tax", T.D152), ("l_returnflag", "Utf8"), ("l_linestatus", "Utf8"), ("l_shipdate", "Date32")]]
src = g.MemoryExec([g.DeviceTable(cols, 0)])
node = T.q1_plan(src)
while not (isinstance(node, g.AggregateExec) and node.mode == "Partial"):
    node = node.children()[0]
s_, pred, m = g.plan._fuse(node.input)
text = g.compile_jit_source(node._descriptor(src.schema(), pred, m), 3)
open(os.path.join(out, "q1.hip"), "w").write(text)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-x", "hip", os.path.join(out, "q1.hip"),
                "-I", os.path.join(ROOT, "arrow-ballista_amd", "csrc"), "-o", os.path.join(out, "q1.s")], check=True, capture_output=True)
asm = open(os.path.join(out, "q1.s")).read()
import re
for k in (".vgpr_count", ".sgpr_count", ".group_segment_fixed_size", ".private_segment_fixed_size"):
    print(k, re.findall(re.escape(k) + r":\s*(\d+)", asm))
