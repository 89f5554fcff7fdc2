# Diagnostic: action-chunk replay (PointTSP-25, K = 2048, fresh rows) for each variant library under lib/variants/.
for so in combinatorial-rl-tasks_amd/lib/variants/*.so; do echo "== $so"; ZENV_LIB_PATH=$so timeout -k 10 200 python scripts/chunk_replay_diag.py 2>&1 | grep -E "K=2048 back|scripted greedy 2048 again"; done
echo "== shipped"; timeout -k 10 200 python scripts/chunk_replay_diag.py 2>&1 | grep -E "K=2048 back|again"
