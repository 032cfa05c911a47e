"""kaldi-lstm_amd: MI355X-native LstmProjectedStreams forward/BPTT/update engine.

The product is the C-ABI shared library `libklstm.so` (include/klstm.h, hand-written gfx950
HIP kernels) plus the C++ mirror of the reference component (include/klstm_component.hpp).
This Python package is plumbing over that ABI for tests, bench.py and torch.distributed
launch; it contains no arithmetic and NO CPU fallback: if the library or a GPU is missing,
calls raise.
"""
from .binding import (Engine, KlstmError, RcclComm, OneshotAllreduce, lib_path, load_library, time_shift, affine_propagate,  # noqa: F401
                      affine_backpropagate, affine_update, affine_gradient, sgd_momentum_update, softmax,
                      xent_eval_masked, xent_eval_masked_post, softmax_xent_masked, debug_gemm_bf16_nt2, DEFER_MOMENTUM)
from .batcher import MultiStreamBatcher  # noqa: F401,E402
from .dp import (DataParallelLstm, DataParallelNnet, LstmDP, AffineDP, SoftmaxXentDP,  # noqa: F401,E402
                 shard_time_major)
