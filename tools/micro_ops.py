import sys, os, time, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch
import arrow_ballista_amd as g, tpch_util as T
from arrow_ballista_amd.expr import col, lit, binary, Operator as Op
tc=g.TaskContext(device=0)
n=1<<27
li=T.gen_lineitem_device(tc,n,columns=("l_orderkey","l_extendedprice","l_discount","l_shipdate"))
s=li.schema()
def timeit(f,reps=3):
    best=None
    for i in range(reps+1):
        tc.sync(); t0=time.perf_counter(); r=f(); tc.sync(); dt=time.perf_counter()-t0
        if i>0: best=dt if best is None or dt<best else best
    return best*1e3, r
# sort by l_extendedprice (Decimal128(15,2), 16-B key; config #5 shape)
plan=g.SortExec([{"expr":col("l_extendedprice",s),"asc":True,"nulls_first":False}], g.MemoryExec([li]))
ms,r=timeit(lambda: plan.execute(0,tc))
print("sort %d rows by l_extendedprice: %.2f ms  %.2f Grows/s  floor(N*(16+8)*2)=%.1f GB -> %.0f GB/s"%(n,ms,n/ms/1e6,n*48/1e9,n*48/ms/1e6))
plan2=g.SortExec([{"expr":col("l_orderkey",s),"asc":False,"nulls_first":False},{"expr":col("l_shipdate",s),"asc":True,"nulls_first":False}], g.MemoryExec([li]))
ms,r=timeit(lambda: plan2.execute(0,tc))
print("sort %d rows by (orderkey desc, shipdate): %.2f ms  %.2f Grows/s"%(n,ms,n/ms/1e6))
for np_ in (8,16,256):
    rp=g.RepartitionExec(g.MemoryExec([li]),[col("l_orderkey",s)],np_)
    ms,r=timeit(lambda: rp.execute_all(0,tc))
    print("partition %d rows into %d: %.2f ms  %.2f Grows/s (perm only; N*(8 read + 4 write))"%(n,np_,ms,n/ms/1e6))
# high-cardinality aggregate: sum(ext*(1-disc)) group by l_orderkey  (n/4 groups)
rev=binary(col("l_extendedprice",s),Op.Multiply,binary(lit(1,("Decimal128",20,0)),Op.Minus,col("l_discount",s)))
agg=g.AggregateExec("Single",[(col("l_orderkey",s),"k")],[{"fn":"SUM","expr":rev,"name":"rev"}],g.MemoryExec([li]),strategy="hash",expected_groups=n//4)
ms,r=timeit(lambda: agg.execute(0,tc))
print("hash aggregate %d rows -> %d groups: %.2f ms  %.2f Grows/s"%(n,r.num_rows,ms,n/ms/1e6))
agg2=g.AggregateExec("Single",[(col("l_orderkey",s),"k")],[{"fn":"SUM","expr":rev,"name":"rev"}],g.MemoryExec([li]),strategy="radix",expected_groups=n//4)
ms,r=timeit(lambda: agg2.execute(0,tc))
print("radix aggregate %d rows -> %d groups: %.2f ms  %.2f Grows/s"%(n,r.num_rows,ms,n/ms/1e6))
