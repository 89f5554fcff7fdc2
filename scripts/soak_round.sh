#!/bin/bash
# The round's parity soak (run on the GPU box through gpurun): persistent, per-step, sliced / one-launch / ragged slices,
# action chunks.
out=${1:-gpurun_out/parity_soak.log}
(
echo "# python tests/soak_parity.py 65536 3000 persistent"; python tests/soak_parity.py 65536 3000 persistent; echo
echo "# python tests/soak_parity.py 65536 3000 per_step"; python tests/soak_parity.py 65536 3000 per_step; echo
echo "# python tests/soak_parity.py 1048576 300 persistent   (slices of 65536, the default)"; python tests/soak_parity.py 1048576 300 persistent; echo
echo "# python tests/soak_parity.py 1048576 300 persistent 0   (one launch over the batch)"; python tests/soak_parity.py 1048576 300 persistent 0; echo
echo "# python tests/soak_parity.py 300000 300 persistent 100000   (ragged slices)"; python tests/soak_parity.py 300000 300 persistent 100000; echo
echo "# python tests/soak_parity.py 65536 2048 chunk   (the recorded actions replayed through zenv_step_many)"; python tests/soak_parity.py 65536 2048 chunk
) > $out 2>&1
tail -3 $out; echo "bit-identical lines: $(grep -c 'bit-identical  (' $out), mismatches: $(grep -c MISMATCH $out)"
