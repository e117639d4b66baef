"""Where does one SortExec.execute() spend its wall time?  (cProfile of the host side + per-phase timers.)"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tpch_util as T
import arrow_ballista_amd as g
from arrow_ballista_amd.expr import col
tc = g.TaskContext(device=0)
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
li = T.gen_lineitem_device(tc, n, columns=("l_orderkey", "l_extendedprice", "l_shipdate"))
src = g.MemoryExec([li]); s = src.schema()
plan = g.SortExec([{"expr": col("l_extendedprice", s), "asc": True, "nulls_first": False}], src)
for _ in range(3):
    plan.execute(0, tc)
tc.ctx.jit_wait(); tc.sync()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); v = plan.execute(0, tc); t1 = time.perf_counter(); tc.sync(); t2 = time.perf_counter()
pr.disable()
print("execute returned after %.2f ms, sync after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
