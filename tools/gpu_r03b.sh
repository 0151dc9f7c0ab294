#!/bin/bash
# round-3 second GPU call: new boundary / group tests, then traversal diagnostics for the greedy and the SAH-optimal 8-wide collapse
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_boundary.py tests/test_group.py tests/test_cpp_host_mirror.py tests/test_materials.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout -k 10 300 python3 tools/visit_probe.py '{"wide_collapse": 0}' '{"wide_collapse": 1, "wide_cost_tri": 0.3}' '{"wide_collapse": 1, "wide_cost_tri": 1.0}' > $O/visit.log 2>&1; cat $O/visit.log
for o in '{"wide_collapse": 0}' '{"wide_collapse": 1, "wide_cost_tri": 0.3}' '{"wide_collapse": 1, "wide_cost_tri": 1.0}'; do timeout -k 10 200 python3 tools/stream_probe.py "$o" >> $O/stream.log 2>&1; done; cat $O/stream.log
