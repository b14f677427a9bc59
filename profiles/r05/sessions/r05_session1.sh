#!/bin/bash
# round 5, GPU session 1: measurements that need no new code — the f64 MFMA co-issue probe (review item 6) and the NEAR-pass
# policy at small sizes (review item 5).  Outputs under gpurun_out/r05/.
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 120 tools/micro/mfma_f64_coissue 40 > $O/mfma_f64_coissue.log 2>&1 || { echo "mfma probe failed"; tail -5 $O/mfma_f64_coissue.log; exit 1; }
cat $O/mfma_f64_coissue.log
# NEAR policy at 1024² (1.05 M rays) and at the N = 8 share of 4096² (2.1 M rays), interleaved A/B
S="rounds=2;waves_per_cu_near=8;waves_per_cu_near=12;waves_per_cu_near=2;qchunk_near=16;qchunk_near=32;qchunk_near=1;near_early=32;near_early=128"
timeout -k 10 500 python tools/launch_ab.py ab --size 1024 --variants ks_ref0,ks_true08 --rounds 4 --sets "$S" > $O/near_policy_1024.log 2>&1 || { echo "ab 1024 failed"; tail -5 $O/near_policy_1024.log; exit 1; }
cat $O/near_policy_1024.log
timeout -k 10 500 python tools/launch_ab.py ab --size 4096 --shares 8 --variants ks_ref0,ks_true08 --rounds 4 --sets "$S" > $O/near_policy_share8.log 2>&1 || { echo "ab share8 failed"; tail -5 $O/near_policy_share8.log; exit 1; }
cat $O/near_policy_share8.log
