import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import arrow_ballista_amd as g, tpch_util as T
tc=g.TaskContext(device=0)
for n in (1,5):
    li=T.gen_lineitem_device(tc,n,seed=7)
    print('oracle',T.q1_oracle_rows(n,seed=7))
    print('single',T.q1_result_to_rows(tc,T.run_q1(tc,li,two_phase=False)))
    print('single-hash',T.q1_result_to_rows(tc,T.run_q1(tc,li,two_phase=False,strategy='hash')))
    plan=T.q1_plan(g.MemoryExec([li]),True)
    # dig out partial
    p=plan
    while not (isinstance(p,g.AggregateExec) and p.mode=='Partial'): p=p.children()[0]
    part=p.execute(0,tc)
    print('partial',T.table_to_rows(tc,part))
