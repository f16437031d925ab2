mkdir -p gpurun_out/r06_s32
MEMC_RANDOM_SEED=777002 MEMC_RANDOM_CASES=600 MEMC_STRIDED_CASES=600 timeout 1200 python -m pytest tests -q -m gpu -k "random_strided_views and 3x9x90x130-smooth-40" -p no:cacheprovider 2>&1 | tail -120 > gpurun_out/r06_s32/fail.log
MEMC_RANDOM_SEED=777002 MEMC_RANDOM_CASES=600 MEMC_STRIDED_CASES=600 timeout 2400 python -m pytest tests -q -m gpu -k "random_strided_views" -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r06_s32/all.log
MEMC_RANDOM_SEED=777002 MEMC_RANDOM_CASES=600 timeout 1200 python -m pytest tests -q -m gpu -k "random_shapes_every_operator and 3x9x90x130-smooth-40" -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r06_s32/contig.log
