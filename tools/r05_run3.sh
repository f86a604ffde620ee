#!/bin/bash
# round 5, GPU call 3: unconditional stores (WHOLE / WT instantiations) against the predicated kernels, interleaved on one lease; parity of the new instantiations
export TMPDIR=/tmp
O=gpurun_out/r05_run3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "G2_full_size or G1_full_size or benched_instantiations or full_size_properties or xna_mfma or slide or golden or single_call" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
bash tools/ab_lib.sh tools/bin/libnaf_nowt.so "G2-k15 G2-k11 G2-k7 G1 G3 G4" 3 100 2>&1 | tee $O/ab_whole_tiles.txt
