"""Writes examples/q1_plan.json: TPC-H q1 as the JSON plan the native executor takes (no GPU needed)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import arrow_ballista_amd as g, tpch_util as T
from arrow_ballista_amd.native import plan_to_json
cols = [g.DeviceColumn(n, t, None, 0, nullable=False) for n, t in [("l_quantity", T.D152), ("l_extendedprice", T.D152), ("l_discount", T.D152), ("l_tax", T.D152),
                                                                   ("l_returnflag", "Utf8"), ("l_linestatus", "Utf8"), ("l_shipdate", "Date32")]]
plan = T.q1_plan(g.MemoryExec([g.DeviceTable(cols, 0)]), two_phase=True)
inputs = []
j = plan_to_json(plan, None, inputs)
open(os.path.join(ROOT, "examples", "q1_plan.json"), "w").write(json.dumps(j, indent=1) + "\n")
print("wrote examples/q1_plan.json,", len(inputs), "input slot(s)")
