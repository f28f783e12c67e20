"""GPU box: the bf16 detector forward of 32 images as one launch sequence against two halves of 16 on two streams (do the
latency-bound small-map layers of one half hide behind the other half's?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

def main():
    dev = torch.device("cuda:0")
    cfg, contract, _, scene = bench.DET_VARIANTS["mbv3s"]
    model = bench.build_and_sync_weights(cfg, contract, dev, 0, 1, scene=scene)
    model.set_compute_dtype("bf16")
    x = torch.randn(32, 3, 736, 1280, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def one():
        with torch.no_grad():
            return model(x)["maps"]
    def two(k=2):
        cur = torch.cuda.current_stream()
        outs = []
        ss = [s1, s2]
        for i in range(k):
            ss[i].wait_stream(cur)
        with torch.no_grad():
            for i in range(k):
                with torch.cuda.stream(ss[i]):
                    outs.append(model(x[i * 32 // k:(i + 1) * 32 // k])["maps"])
        for i in range(k):
            cur.wait_stream(ss[i])
        return outs
    for name, fn in (("one stream, 32", one), ("two streams, 16 + 16", two), ("one stream, 32", one), ("two streams, 16 + 16", two)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): fn()
        torch.cuda.synchronize()
        print("%-24s %.3f ms per 32 images" % (name, (time.perf_counter() - t0) / 30 * 1e3))
    a = one(); b = torch.cat(two(), 0)
    torch.cuda.synchronize()
    print("identical:", torch.equal(a, b))

main()
