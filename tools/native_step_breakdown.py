"""Where a bench.py step goes on the native path: partial stage, final stage, per-node metrics, host-side share."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import arrow_ballista_amd as g, tpch_util as T
tc = g.TaskContext(device=0)
n = T.LINEITEM_ROWS[10]
li = T.gen_lineitem_device(tc, n)
partial_py, full_py, final_src = T.q1_split_plan(li, 64)
partial = g.NativePlan(partial_py, tc)
res0 = partial.execute(0)
final_src.partitions[0] = res0.to_device_table(tc.device)
final = g.NativePlan(full_py, tc)
import gc; gc.collect(); gc.freeze(); gc.disable()
K = 50
def loop(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
def both():
    r = partial.execute(0); final.set_input_result(0, r); return final.execute(0)
res = partial.execute(0); final.set_input_result(0, res)
print("partial stage  %.3f ms" % loop(lambda: partial.execute(0)))
print("final stage    %.3f ms" % loop(lambda: final.execute(0)))
print("both           %.3f ms" % loop(both))
partial.profile(True); loop(lambda: partial.execute(0)); ms, launches, desc = partial.profile(False)
print("dominant kernel %.3f ms avg over %d launches" % (ms / max(1, launches), launches))
for name, pl in (("partial", partial), ("final", final)):
    m0 = pl.metrics(); loop(lambda: pl.execute(0)); m1 = pl.metrics()
    for a, b in zip(m0, m1):
        print("  %-8s %-24s %8.1f us/step" % (name, b["node"], (b["elapsed_compute"] - a["elapsed_compute"]) / (K + 5) / 1e3))
