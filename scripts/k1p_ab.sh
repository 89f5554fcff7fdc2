# Diagnostic: same-box A/B of the persistent kernel between the shipped library and variant libraries under lib/variants/
# (scripts/k1p_rate.py through ZENV_LIB_PATH), A B A B.
for rep in 1 2; do
  echo "== shipped"; timeout -k 10 200 python scripts/k1p_rate.py
  for so in combinatorial-rl-tasks_amd/lib/variants/*.so; do echo "== $so"; ZENV_LIB_PATH=$PWD/$so timeout -k 10 200 python scripts/k1p_rate.py; done
done
