for e in DN_X=1 DN_NO_X3_SPLITK=1 DN_REDUCE_ROWS_PER_THREAD=8 DN_NO_WINO_SPLITK=1 "DN_NO_X3_SPLITK=1 DN_REDUCE_ROWS_PER_THREAD=8 DN_NO_WINO_SPLITK=1" DN_NO_X3_DIRECT=1 DN_COMPUTE=f32; do
echo "== $e"; env $e python -m pytest tests/test_gpu_models.py -q -m gpu -s -k "fcrn_aspp" 2>&1 | grep "whole-gradient\|passed\|failed"
done
