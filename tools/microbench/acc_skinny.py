import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools", "microbench"))
import gemm_accounting as G
M = 204800
for N, K, oa, ob, tag in [(192, 64, 0, 1, "LC qkv fwd"), (64, 64, 0, 1, "LC unify fwd"), (256, 64, 0, 1, "LC ff1 fwd"), (64, 256, 0, 1, "LC ff2 fwd"),
                          (256, 64, 0, 0, "LC ff2 dgrad"), (64, 256, 0, 0, "LC ff1 dgrad"), (64, 192, 0, 0, "LC qkv dgrad")]:
    G.account(tag, M, N, K, oa, ob)
for Mw, Nw, tag in [(192, 64, "LC qkv wgrad"), (256, 64, "LC ff1 wgrad"), (64, 256, "LC ff2 wgrad"), (64, 64, "LC unify wgrad")]:
    G.account(tag, Mw, Nw, M, 1, 0)
