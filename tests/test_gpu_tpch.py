"""TPC-H q3 / q5 operator pipelines on the device (reference benchmarks/queries/q3.sql, q5.sql) against
the oracle on the same seeded synthetic tables (generator restated on the CPU by oracle/oracle.c)."""
import pytest

import arrow_ballista_amd as g
import tpch_util as T

pytestmark = pytest.mark.gpu


def _tables(tc, n_li, n_cust, n_supp):
    n_orders = (n_li + 3) // 4
    cols = ("l_orderkey", "l_suppkey", "l_extendedprice", "l_discount", "l_shipdate")
    li = T.gen_lineitem_device(tc, n_li, n_supp=n_supp, columns=cols)
    od = T.gen_orders_device(tc, n_orders, n_cust)
    cu = T.gen_customer_device(tc, n_cust)
    su = T.gen_supplier_device(tc, n_supp)
    host_li = T.lineitem_host_to_arrow(T.gen_lineitem_host(n_li, n_supp=n_supp), n_li)
    h_or, h_cu, h_su = T.gen_other_tables_host(n_orders, n_cust, n_supp)
    # device generators == oracle restatement
    for dev, host in ((od, h_or), (cu, h_cu), (su, h_su)):
        assert dev.to_arrow(tc.ctx).to_pydict() == {k: v for k, v in host.to_pydict().items()}
    return (li, od, cu, su), (host_li, h_or, h_cu, h_su)


@pytest.mark.parametrize("n_li", [4000, 120_000])
def test_q3(tc, n_li):
    (li, od, cu, su), (hl, ho, hc, hs) = _tables(tc, n_li, 1500, 100)
    plan = T.q3_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li]))
    got = T.table_to_rows(tc, g.plan.materialize(tc, plan.execute(0, tc)))
    exp = T.q3_oracle(hc, ho, hl)
    assert len(got) == len(exp) and len(exp) > 0
    assert [(r[1], r[2]) for r in got] == [(r[1], r[2]) for r in exp]          # ORDER BY revenue desc, o_orderdate
    assert sorted(got) == sorted(exp)


@pytest.mark.parametrize("n_li", [4000, 120_000])
def test_q5(tc, n_li):
    import pyarrow as pa
    (li, od, cu, su), (hl, ho, hc, hs) = _tables(tc, n_li, 1500, 100)
    nation, region = T.nation_region_arrow()
    plan = T.q5_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li]), g.MemoryExec([su]), g.MemoryExec([nation]), g.MemoryExec([region]))
    got = T.table_to_rows(tc, g.plan.materialize(tc, plan.execute(0, tc)))
    exp = T.q5_oracle(hc, ho, hl, hs)
    assert [tuple(r) for r in got] == [tuple(r) for r in exp] and len(exp) > 0
