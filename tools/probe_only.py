import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import arrow_ballista_amd as g, tpch_util as T, bench_extras as X
tc=g.TaskContext(device=0)
print(X.join_probe_micro(tc,g,24,28,1.0,reps=2))
