import sys, os, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch
import arrow_ballista_amd as g, tpch_util as T
from arrow_ballista_amd.expr import col, lit, binary, and_, Operator as Op
tc=g.TaskContext(device=0)
n=T.LINEITEM_ROWS[10]
li=T.gen_lineitem_device(tc,n)
s=li.schema()
D=("Decimal128",15,2)
def time_filter(name,pred):
    op=tc.op({"op":"filter","input":{"fields":s},"predicate":pred})
    inp,k=li.input_struct()
    cnt=torch.zeros(2,dtype=torch.int64,device=tc.device)
    for i in range(3):
        op.profile(True)
        tc.ctx.check(tc.ctx.L.gpuq_filter_run(op.h,tc.stream_ptr(),C.byref(inp),0,None,cnt.data_ptr()))
        tc.sync(); ms,_=op.profile(False)
    print("%-40s %.3f ms"%(name,ms))
def time_agg(name,groups,aggs,pred=None,strategy="tiny"):
    d={"op":"aggregate","mode":"Partial","input":{"fields":s},"strategy":strategy,"group_expr":[{"expr":e,"name":n_} for e,n_ in groups],"aggr_expr":aggs}
    if pred is not None: d["predicate"]=pred
    for i in range(3):
        op=tc.op(d); op.profile(True)
        out=g.plan.aggregate_table(tc,li,d)
        ms,_=op.profile(False)
    print("%-40s %.3f ms  groups=%d"%(name,ms,out.num_rows))
ship=binary(col("l_shipdate",s),Op.LtEq,lit(T.Q1_SHIPDATE_MAX,"Date32"))
time_filter("filter shipdate (1 col, 4B/row)",ship)
time_filter("filter qty>0 (1 col 16B/row)",binary(col("l_quantity",s),Op.Gt,lit(0,D)))
time_filter("filter 4 decimals (64B/row)",and_(binary(col("l_quantity",s),Op.Gt,lit(0,D)),binary(col("l_extendedprice",s),Op.Gt,lit(0,D)),binary(col("l_discount",s),Op.GtEq,lit(0,D)),binary(col("l_tax",s),Op.GtEq,lit(0,D))))
time_filter("filter rf='A' (utf8)",binary(col("l_returnflag",s),Op.Eq,lit("A")))
time_filter("filter rf,ls (2 utf8)",and_(binary(col("l_returnflag",s),Op.Eq,lit("A")),binary(col("l_linestatus",s),Op.Eq,lit("F"))))
cnt=[{"fn":"COUNT","expr":lit(1),"name":"c"}]
time_agg("agg count(*) no group",[],cnt)
time_agg("agg count(*) no group + pred",[],cnt,ship)
time_agg("agg sum(qty) no group",[],[{"fn":"SUM","expr":col("l_quantity",s),"name":"s"}])
time_agg("agg count by rf",[(col("l_returnflag",s),"rf")],cnt)
time_agg("agg count by rf,ls",[(col("l_returnflag",s),"rf"),(col("l_linestatus",s),"ls")],cnt)
time_agg("agg sum(qty),sum(ext) by rf,ls",[(col("l_returnflag",s),"rf"),(col("l_linestatus",s),"ls")],[{"fn":"SUM","expr":col("l_quantity",s),"name":"a"},{"fn":"SUM","expr":col("l_extendedprice",s),"name":"b"}])
