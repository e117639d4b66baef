import sys, os, time, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch
import arrow_ballista_amd as g, tpch_util as T
tc=g.TaskContext(device=0)
sf=100
n_li=T.LINEITEM_ROWS[sf]; n_orders,n_cust,n_supp=(n_li+3)//4,int(150000*sf),int(10000*sf)
li=T.gen_lineitem_device(tc,n_li,n_supp=n_supp,columns=("l_orderkey","l_suppkey","l_extendedprice","l_discount","l_shipdate"))
od=T.gen_orders_device(tc,n_orders,n_cust); cu=T.gen_customer_device(tc,n_cust); su=T.gen_supplier_device(tc,n_supp)
nation,region=T.nation_region_arrow()
for name,mk in (("q3",lambda:T.q3_plan(g.MemoryExec([cu]),g.MemoryExec([od]),g.MemoryExec([li]))),
                ("q5",lambda:T.q5_plan(g.MemoryExec([cu]),g.MemoryExec([od]),g.MemoryExec([li]),g.MemoryExec([su]),g.MemoryExec([nation]),g.MemoryExec([region])))):
    p=g.NativePlan(mk(),tc)
    for i in range(3): r=p.execute(0)
    m0=p.metrics()
    tc.sync(); t0=time.perf_counter(); r=p.execute(0); tc.sync(); dt=time.perf_counter()-t0
    m1=p.metrics()
    print(name,"%.2f ms"%(dt*1e3))
    for a,b in zip(m0,m1): print("   %-22s rows=%-12d own=%.3f ms"%(a["node"],b["output_rows"]-a["output_rows"],(b["elapsed_compute"]-a["elapsed_compute"])/1e6))
