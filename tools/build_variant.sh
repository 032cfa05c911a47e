#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." : a second libklstm (kaldi-lstm_amd/libklstm_NAME.so) built with extra defines, for A-B
# runs with KLSTM_LIB_PATH=... (binding.py).  Objects under kaldi-lstm_amd/build_NAME/.
set -e
cd "$(dirname "$0")/../kaldi-lstm_amd"
name=$1; shift
mkdir -p build_$name
for f in klstm_kernels klstm_persist klstm_persist_bwd klstm_persist_ms klstm_persist_xl klstm_fold klstm_fold3 klstm_oneshot klstm_outer klstm_gemm16 klstm_engine; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -I/opt/rocm/include "$@" -c csrc/$f.hip -o build_$name/$f.o &
done
wait; for f in klstm_kernels klstm_engine klstm_fold3; do test -f build_$name/$f.o || { echo "compile of $f failed"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libklstm_$name.so build_$name/*.o -ldl
echo kaldi-lstm_amd/libklstm_$name.so
