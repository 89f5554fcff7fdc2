#!/bin/bash
# CPU only: the oracle (gcc) rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer in a scratch copy of the tree,
# then every CPU test that drives it (single envs, the OpenMP batch driver, the gloo shards).  GPU sanitizers are not
# available on this pool; the kernels mirror the oracle's arithmetic token for token, so an out-of-range shift or a
# signed overflow there would show here.   usage: scripts/sanitize_oracle.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/zenv_san.XXXX)
(cd "$root" && git archive HEAD) | tar -x -C "$tmp"
mkdir -p "$tmp/combinatorial-rl-tasks_amd/lib"
cp "$root/combinatorial-rl-tasks_amd/lib/libzenv_hip.so" "$tmp/combinatorial-rl-tasks_amd/lib/"
sed -i 's/cmd = \["gcc", "-O2",/cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",/' "$tmp/oracle/oracle.py"
cd "$tmp"
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 \
python -m pytest tests/test_oracle_cpu.py tests/test_dynamics_independent.py tests/test_hard_env_and_exception.py \
    tests/test_sharding_gloo.py -x -q -m "not gpu" -p no:cacheprovider
nm -D oracle/build/libzenv_oracle.so | grep -q __asan && echo "oracle was built with the sanitizers: clean"
rm -rf "$tmp"
