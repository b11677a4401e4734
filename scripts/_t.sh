timeout 900 python -m pytest tests/test_infer.py -x -q -m gpu 2>&1 | tail -3
export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_fdg -o p -- python3 scripts/fd_graph_timeline.py run > /dev/null 2>&1; python3 scripts/fd_graph_timeline.py report | tee gpurun_out/fd_graph_timeline.txt; rm -rf gpurun_out/prof_fdg
