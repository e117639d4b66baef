import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
import arrow_ballista_amd as g, tpch_util as T
tc=g.TaskContext(device=0)
n=1000
cols=("l_orderkey","l_suppkey","l_quantity","l_extendedprice","l_discount","l_tax","l_returnflag","l_linestatus","l_shipdate")
li=T.gen_lineitem_device(tc,n,seed=7,columns=cols)
host=T.gen_lineitem_host(n,seed=7)
for c in li.columns:
    d=c.data.cpu().numpy()
    h=host[c.name].view(np.uint8)
    nb=min(len(h), len(d))
    bad=np.nonzero(d[:nb]!=h[:nb])[0]
    print(c.name, 'mismatching bytes:', len(bad), bad[:10].tolist())
