cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04; mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_chunk -- python scripts/chunk_pmc.py > $out/chunk_pmc.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_chunk -- python scripts/chunk_pmc.py >> $out/chunk_pmc.log 2>&1
tail -2 $out/chunk_pmc.log
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json
