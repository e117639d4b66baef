"""Two host threads drive ONE device through ONE gpuq context, each on its own HIP stream -- how the reference's executor runs
`concurrent_tasks` tasks on its task-runner pool (ballista/executor/src/cpu_bound_executor.rs:94-131,
executor_server.rs:1027-1032).  The library's workspaces, join tables and results are recycled through a process-wide pool;
a block released under one stream and taken under another is ordered by an event (csrc/devbuf.h).  Every result of every
iteration is checked against the oracle."""
import threading

import pytest

import arrow_ballista_amd as g
import tpch_util as T

pytestmark = pytest.mark.gpu


def test_two_threads_two_streams_one_context(tc):
    import torch
    dev = tc.device
    errors, done = [], []

    def worker(wid, sizes):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                mytc = g.TaskContext(ctx=tc.ctx)
                assert mytc.stream_ptr().value == stream.cuda_stream
                for it, n_li in enumerate(sizes):
                    # q3: two join tables, pair vectors, a hash aggregate and a sort per run -- all pooled allocations of
                    # different sizes per iteration, so blocks migrate between the two threads' streams
                    n_cust = 1500
                    cols = ("l_orderkey", "l_suppkey", "l_extendedprice", "l_discount", "l_shipdate")
                    li = T.gen_lineitem_device(mytc, n_li, n_supp=100, columns=cols)
                    od = T.gen_orders_device(mytc, (n_li + 3) // 4, n_cust)
                    cu = T.gen_customer_device(mytc, n_cust)
                    plan = g.NativePlan(T.q3_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li])), mytc)
                    got = None
                    for _ in range(3):
                        res = plan.execute(0)
                        got = res
                    rows = [tuple(r) for r in __import__("test_gpu_native_plan").arrow_rows(got.to_arrow())]
                    h = T.gen_q3_tables_host(n_li, n_cust)
                    exp, _st = T.q3_oracle_c(h)
                    assert sorted(rows) == sorted(exp), "thread %d iteration %d (%d rows): q3 differs from the oracle" % (wid, it, n_li)
                    # q1 on the same stream: LDS aggregate workspaces
                    l1 = T.gen_lineitem_device(mytc, n_li // 2 + 17, seed=5 + wid)
                    assert T.q1_result_to_rows(mytc, T.run_q1(mytc, l1)) == T.q1_oracle_rows(n_li // 2 + 17, seed=5 + wid)
                stream.synchronize()
            done.append(wid)
        except BaseException as e:      # noqa: BLE001 -- reported by the main thread
            errors.append((wid, repr(e)))

    a = threading.Thread(target=worker, args=(0, [40_000, 200_000, 12_000, 90_000, 300_000, 64_000]))
    b = threading.Thread(target=worker, args=(1, [150_000, 9_000, 260_000, 33_000, 70_000, 210_000]))
    a.start(); b.start(); a.join(); b.join()
    assert not errors, errors
    assert sorted(done) == [0, 1]
