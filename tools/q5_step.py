"""SF100 q5 through the native executor, 6 executions (for a rocprofv3 kernel trace: tools/gpu_r03i.sh) + per-operator times."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import arrow_ballista_amd as g
from benchmarks import tpch as T
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 100
tc = g.TaskContext(device=0)
tc.ctx.set_jit("wait")
n_li = T.LINEITEM_ROWS.get(int(sf), int(6_000_000 * sf))
n_orders, n_cust, n_supp = (n_li + 3) // 4, int(150_000 * sf), int(10_000 * sf)
li = T.gen_lineitem_device(tc, n_li, n_supp=n_supp, columns=("l_orderkey", "l_suppkey", "l_extendedprice", "l_discount", "l_shipdate"))
od = T.gen_orders_device(tc, n_orders, n_cust); cu = T.gen_customer_device(tc, n_cust); su = T.gen_supplier_device(tc, n_supp)
nation, region = T.nation_region_arrow()
plan = g.NativePlan(T.q5_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li]), g.MemoryExec([su]), g.MemoryExec([nation]), g.MemoryExec([region])), tc)
ts = []
for r in range(7):
    if r == 2:
        plan.profile(True)
    tc.sync(); t0 = time.perf_counter(); res = plan.execute(0); tc.sync(); ts.append((time.perf_counter() - t0) * 1e3)
ops = plan.profile_all()
import re
print("q5 SF%g ms per execution:" % sf, [round(t, 2) for t in ts], plan.exec_stats(), "rows", res.num_rows)
for o in sorted(ops, key=lambda o: -o.get("op_ms", 0)):
    m = re.search(r'"label":"(\w+)"', o.get("desc", ""))
    if o.get("op_ms", 0) > 0:
        print("  %-24s op_ms %.3f kernel_ms %.3f x%d" % ((m.group(1) if m else o["op"]), o["op_ms"] / 5, o["kernel_ms"] / 5, o["launches"] // 5))
