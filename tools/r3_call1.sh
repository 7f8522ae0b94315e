#!/bin/bash
# round 3, GPU call 1: instruction issue cost table, FETCH_SIZE calibration on gather patterns, baseline A/B line
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
$R/tools/bin/issue_test > $O/r3_issue_test.log 2>&1
$R/tools/bin/gather_calib > $O/r3_gather_plain.log 2>&1
rocprofv3 -L > $O/r3_counters_list.txt 2>&1
timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE -d $O/r3_gather_fetch -o f --output-format csv -- $R/tools/bin/gather_calib > $O/r3_gather_fetch.log 2>&1
timeout -k 5 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/r3_gather_rdreq -o f --output-format csv -- $R/tools/bin/gather_calib > $O/r3_gather_rdreq.log 2>&1
timeout -k 5 120 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/r3_gather_tcc -o f --output-format csv -- $R/tools/bin/gather_calib > $O/r3_gather_tcc.log 2>&1
cd $R && AB_TILES=10 timeout 300 python tools/abbench.py libzxc_mi355x.so > $O/r3_ab_base.log 2>&1
tail -3 $O/r3_ab_base.log; cat $O/r3_issue_test.log | head -40; cat $O/r3_gather_plain.log
