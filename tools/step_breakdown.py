import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch
import arrow_ballista_amd as g, tpch_util as T
tc=g.TaskContext(device=0)
n=T.LINEITEM_ROWS[10]
li=T.gen_lineitem_device(tc,n)
full=T.q1_plan(g.MemoryExec([li]),two_phase=True)
def walk(p,d=0,out=None):
    out=[] if out is None else out
    out.append((d,p))
    for c in p.children(): walk(c,d+1,out)
    return out
for i in range(3): g.plan.materialize(tc, full.execute(0,tc))
import gc; gc.collect(); gc.freeze(); gc.disable()
nodes=walk(full)
for d,p in nodes: p.metrics.elapsed_compute_ns=0
torch.cuda.synchronize(); t0=time.perf_counter()
K=20
for i in range(K):
    out=g.plan.materialize(tc, full.execute(0,tc))
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/K*1e3
print("ms/step %.3f"%dt)
for d,p in nodes: print("  "*d+"%-22s %8.3f ms (own, excl. children's exec)"%(type(p).__name__+('/'+p.mode if hasattr(p,'mode') else ''), p.metrics.elapsed_compute_ns/K/1e6))
# materialize cost
torch.cuda.synchronize(); t0=time.perf_counter()
for i in range(K): v=full.execute(0,tc)
torch.cuda.synchronize(); print("execute only (no materialize): %.3f ms"%((time.perf_counter()-t0)/K*1e3))
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for i in range(K): out=g.plan.materialize(tc, full.execute(0,tc))
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
