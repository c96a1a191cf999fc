# Round 3, final records: SQ / TCP passes of the three legs under the final tree, then the bench line (reads profiles/bench_pmc*.json).
cd $GRAFT_REPO_ROOT
bash tools/pmc_extra.sh r03 "1 2 4" headline dragon trimmed
