import os, sys, subprocess
sys.path.insert(0, "/root/repo")
mode = sys.argv[1] if len(sys.argv) > 1 else "run"
if mode == "run":
    outs = {}
    envs = {"head": dict(ZAMD_GPU_LIB="/root/repo/build/variants/headlib.so"), "tree-unfused": dict(ZGPU_WALK_FUSE="0"), "tree-fused": dict(ZGPU_WALK_FUSE="1")}
    for k, e in envs.items():
        outs[k] = subprocess.check_output([sys.executable, __file__, "child"], env=dict(os.environ, **e)).decode().splitlines()
    for i, ln in enumerate(outs["head"]):
        print(ln)
        for k in ("tree-unfused", "tree-fused"):
            if outs[k][i] != ln:
                print("   ", k, outs[k][i])
else:
    import hashlib, zlib
    import zlib_amd
    from oracle import cases, corpus_py as CP
    eng = zlib_amd.Engine(0)
    for name, data in (("hello1m", cases.hello_1mib()), ("text", cases.make("text", 70000, 3)), ("text8k", cases.make("text", 8000, 3)), ("text9k", cases.make("text", 9000, 3)), ("zeros", bytes(200000)), ("rand", cases.make("rand", 70000, 5))):
        for lvl in (6,):
            z, offs = eng.deflate_host(data, lvl, want_offsets=True)
            try:
                ok = zlib.decompress(z) == data
            except Exception as e:
                ok = str(e)[:40]
            print(name, lvl, len(z), ok, [hashlib.sha256(z[int(offs[i]):int(offs[i + 1])]).hexdigest()[:8] for i in range(len(offs) - 1)][:4])
