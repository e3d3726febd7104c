#!/bin/bash
# CPU-side sanitizer runs (the GPU pool has no sanitizer builds): the oracle under AddressSanitizer + UBSan through its own
# test files, the TripleBuffer hand-off test under ThreadSanitizer.  Everything is built under /tmp; the tree is untouched.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
W=${TMPDIR:-/tmp}/irmv_sanitize; mkdir -p $W; cd $W
S="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
F="-std=c11 -fPIC -Wall -Wextra -mavx2 -mfma -I $R/oracle"
gcc $F $S -O3 -fopenmp -c $R/oracle/orc_net.c -o orc_net.o
gcc $F $S -O2 -ffp-contract=off -c $R/oracle/orc_post.c -o orc_post.o
gcc $F $S -O2 -ffp-contract=off -c $R/oracle/orc_light.c -o orc_light.o
gcc -shared -fopenmp $S -o liboracle.so orc_net.o orc_post.o orc_light.o -lm
cat > run.py <<PY
import sys
sys.path.insert(0, '$R'); sys.path.insert(0, '$R/tests')
from oracle import oracle
oracle._LIB_PATH = '$W/liboracle.so'
oracle.build = lambda force=False: oracle._LIB_PATH
import pytest
sys.exit(pytest.main(['-x', '-q', '-m', 'not gpu', '-p', 'no:cacheprovider'] + ['$R/tests/' + f for f in (
    'test_oracle_net.py', 'test_oracle_post.py', 'test_oracle_light.py', 'test_oracle_pnp.py', 'test_oracle_preprocess.py',
    'test_shufflenet.py', 'test_int8_weights.py')]))
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
    UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4 python3 run.py
g++ -std=c++20 -O1 -g -fsanitize=thread -pthread -I $R/include $R/tests/cpp/triple_buffer_test.cpp -o tb_tsan
./tb_tsan
echo "sanitizers clean"
