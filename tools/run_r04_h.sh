#!/bin/bash
# Round 4, session H: this tree's records — bench line, rocprofv3 kernel stats and FETCH / WRITE passes per leg (tools/gpu_profile.sh), then the
# SQ / TCP / TA passes (tools/pmc_extra.sh, TA counters one per pass).  Summaries come back through gpurun_out/profiles/.
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
bash tools/gpu_profile.sh r04 2 || { echo "gpu_profile failed"; exit 1; }
bash tools/pmc_extra.sh r04 "1 2 4 3 5 6" headline dragon trimmed || { echo "pmc_extra failed"; exit 1; }
ls gpurun_out/profiles | head -40
