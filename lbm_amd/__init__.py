"""lbm_amd -- MI355X-native latent retrieval for Learning Batteries Market (amazedsaint/lbm).

Drop-in modules for the reference's retrieval hot path, backed by hand-written HIP kernels
(liblbhip.so, C ABI in include/lbhip.h):

    lbm_amd.latent         <->  lb/latent.py          (embed, cosine, _tokenize)
    lbm_amd.context_graph  <->  lb/context_graph.py   (Claim, ContextGraph)
    lbm_amd.node           <->  lb/node.py:718-734    (compile_context, NodeError)
    lbm_amd.sharded        claim-sharded corpus over N GPUs (one process per GPU, RCCL all-gather merge)

Importing the package needs neither a GPU nor the shared library; using it does.
See DESIGN.md and INTEGRATION.md.
"""
__version__ = "0.1.0"
