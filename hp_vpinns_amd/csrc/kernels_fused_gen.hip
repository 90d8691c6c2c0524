// The general variational forms on the whole-iteration kernel (k_iter_fused<.., NT2, GEN>: term weights, trainable epsilon, the mixed
// second tangent) as a translation unit of their own -- the same source, the other half of its instantiations (build time).
#define HPV_FZ_GEN_TU
#include "kernels_fused.hip"
