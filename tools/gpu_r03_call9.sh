#!/usr/bin/env bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
timeout 400 python tools/sweep_small_convs.py > $O/sweep_small.jsonl 2> $O/sweep_small.err; echo "sweep rc=$?"; cat $O/sweep_small.jsonl | cut -c1-260
timeout 400 bash tools/prof_conv.sh 0 > $O/prof_conv_h0.txt 2>&1; tail -12 $O/prof_conv_h0.txt | cut -c1-400
timeout 400 bash tools/prof_conv.sh 6 > $O/prof_conv_h6.txt 2>&1; tail -12 $O/prof_conv_h6.txt | cut -c1-400
