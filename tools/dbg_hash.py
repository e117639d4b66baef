import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch
import arrow_ballista_amd as g, tpch_util as T
for mode in ("off", "force"):
    tc=g.TaskContext(device=0)
    tc.ctx.set_jit(mode)
    for n in (65, 130, 1000):
        li=T.gen_lineitem_device(tc,n,seed=7)
        exp=T.q1_oracle_rows(n,seed=7)
        for tp in (True, False):
            for strat in ("auto","hash","tiny"):
                got=T.q1_result_to_rows(tc,T.run_q1(tc,li,two_phase=tp,strategy=strat))
                print(mode, n, tp, strat, "OK" if got==exp else "DIFF counts got %s exp %s"%([r[-1] for r in got],[r[-1] for r in exp]))
