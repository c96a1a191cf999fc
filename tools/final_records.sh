# Round 3, final records: the whole GPU suite, smoke, the bench line (reads profiles/bench_pmc.json / bench_pmc_extra.json of this tree).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_final_gputests.log 2>&1; echo "exit $?" >> gpurun_out/r03_final_gputests.log)
tail -3 gpurun_out/r03_final_gputests.log
grep -q '^exit 0' gpurun_out/r03_final_gputests.log || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
(timeout -k 10 700 python bench.py > gpurun_out/r03_final_bench_default.log 2>&1; echo "exit $?" >> gpurun_out/r03_final_bench_default.log)
grep '^{"metric"' gpurun_out/r03_final_bench_default.log > gpurun_out/profiles/r03_bench_line.json
tail -1 gpurun_out/r03_final_bench_default.log
