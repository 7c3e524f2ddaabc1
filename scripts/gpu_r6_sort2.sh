export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "onesweep or sort_bit_exact or full_size_sort or bucket_sort or sorted or sort_" 2>&1 | tail -2
for fl in 0 0x100; do for n in 2000000 5000000; do for m in rayon radix_far; do python scripts/sort_rates.py $n $m $fl 2>&1 | grep -v amdgpu.ids; done; done; done
