// The attention (ATT = true) instantiations of egnn_edge_chain_kernel, as a translation unit of their own so that they compile
// beside mdx_egnn_chain.hip instead of behind it (the chain's source is one 2 000-line template; twelve more instantiations in
// the same unit added three minutes to the build).  Same source, nothing else: the C entry points stay in mdx_egnn_chain.hip,
// which reaches these kernels through mdx_chain_launch_attention.
#define MDX_CHAIN_NO_ENTRY_POINTS
#define MDX_CHAIN_ATTENTION_UNIT
#include "mdx_egnn_chain.hip"
