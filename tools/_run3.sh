python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "k_split or conv_family or input_channel_split" 2>&1 | tail -8
bash tools/_sweep.sh "4 8 32" DN_X=1 DN_NO_X3_SPLITK=1 "DN_X3_SPLITK_TARGET=256" "DN_X3_SPLITK_TARGET=1024" "DN_X3_SPLITK_MINCH=4" "DN_REDUCE_ROWS_PER_THREAD=2 DN_REDUCE_MAX_BLOCKS=2048" "DN_REDUCE_ROWS_PER_THREAD=4 DN_REDUCE_MAX_BLOCKS=2048" "DN_REDUCE_ROWS_PER_THREAD=2"
