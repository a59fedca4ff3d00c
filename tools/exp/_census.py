"""shared helper of the launch-census tools: launching aten operations per source line of pyvbmp_amd"""
import collections

import torch  # noqa: F401


def by_source(name, fn, top=40):
    """aten operations that launch, per source line of pyvbmp_amd (python tools/exp/dmbd_launch_census.py --src)"""
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    views = ("view", "reshape", "expand", "squeeze", "unsqueeze", "transpose", "permute", "slice", "select", "alias", "as_strided",
             "detach", "t.default", "size", "stride", "is_", "_unsafe_view", "diagonal", "unbind", "split", "narrow", "empty", "sym_",
             "_local_scalar", "lift_fresh", "unfold", "movedim", "mT", "zeros_like", "new_empty")
    c = collections.Counter()

    class Tr(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            n = str(func)
            if not any(v in n for v in views):
                fr = [f for f in traceback.extract_stack() if "pyvbmp_amd/" in f.filename]
                where = f"{fr[-1].filename.split('pyvbmp_amd/')[-1]}:{fr[-1].lineno}" if fr else "?"
                c[(where, n.replace("aten.", ""))] += 1
            return func(*args, **(kwargs or {}))
    with Tr():
        fn()
    per_line = collections.Counter()
    for (w, n), v in c.items():
        per_line[w] += v
    print(f"== {name}: {sum(c.values())} launching aten operations, by source line")
    for w, v in per_line.most_common(top):
        print(f"   {v:4d} {w}  " + ", ".join(f"{n} x{k}" for (ww, n), k in c.items() if ww == w)[:150])

