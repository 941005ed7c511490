"""GraphedForward under an outer autocast context that is exited (and the allocator cache
emptied) between capture and replay — the situation of the stage-1 query encoder."""
import sys, torch
sys.path.insert(0, "/root/repo")
from tristage_rag_amd.encoders import GraphedForward, load_backbone
for spec in ("random:minilm", "random:bert", "random:modernbert"):
    tok, m, _ = load_backbone(spec, "/tmp/x", "base")
    m = m.cuda().eval()
    gf = GraphedForward(m, 0, torch.bfloat16)
    outs = {}
    for rnd in range(3):
        for n in (5, 12, 30, 7, 60):
            ids = torch.randint(1000, 5000, (1, n), device="cuda", generator=torch.Generator(device="cuda").manual_seed(n))
            mask = torch.ones((1, n), dtype=torch.long, device="cuda")
            with torch.autocast("cuda", dtype=torch.bfloat16):          # outer context, exited every call
                got = gf(ids, mask)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                ref = m(input_ids=ids, attention_mask=mask).last_hidden_state
            err = (got.float() - ref.float()).abs().max().item()
            outs[(rnd, n)] = err
        torch.cuda.empty_cache()                                          # return freed blocks to the driver
        big = torch.empty((1 << 28,), device="cuda"); big.fill_(1.0); del big   # and reuse memory
    torch.cuda.synchronize()
    print(spec, "graphs", sorted(gf._graphs), "broken", gf._broken, "max err %.3e" % max(outs.values()))
