import sys, time, torch
sys.path.insert(0, "/root/repo")
from tristage_rag_amd.encoders import load_backbone
for spec in ("random:bert", "random:modernbert", "random:minilm"):
    tok, m, _ = load_backbone(spec, "/tmp/x", "base")
    m = m.cuda().eval()
    L = 32
    ids = torch.randint(1000, 5000, (1, L), device="cuda")
    mask = torch.ones((1, L), dtype=torch.long, device="cuda"); mask[:, 20:] = 0
    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return m(input_ids=ids, attention_mask=mask).last_hidden_state
    for _ in range(3): ref = fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fwd()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): fwd()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fwd()
        g.replay(); torch.cuda.synchronize()
        err = (out.float() - ref.float()).abs().max().item()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); graphed = (time.perf_counter() - t0) / 20
        print(spec, "eager %.3f ms  graph %.3f ms  maxerr %.2e" % (eager * 1e3, graphed * 1e3, err))
    except Exception as e:
        print(spec, "eager %.3f ms  CAPTURE FAILED: %s" % (eager * 1e3, repr(e)[:300]))
