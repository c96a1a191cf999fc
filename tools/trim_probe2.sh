cd $GRAFT_REPO_ROOT
. tools/trim_probe_fn.sh
FGOICP_TRIM_SAMPLE=0 run "two-pass           "
FGOICP_TRIM_SAMPLE=5 run "one-pass margin 1  "
FGOICP_FINALIZE_SIDE=0 FGOICP_TRIM_SAMPLE=5 run "serial one-pass 1  "
FGOICP_TRIM_SAMPLE=5 FGOICP_TRIM_MARGIN=0.75 run "one-pass margin .75"
