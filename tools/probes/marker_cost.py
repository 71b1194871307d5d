"""What does an event record in the middle of a busy queue cost that queue?  A chain of ~300 us GEMMs on the compute stream, with between
every two of them (a) nothing, (b) an event record, (c) record + a second stream waiting for it and running a tiny kernel, (d) c + the
compute stream waiting for the second stream's event before the NEXT gemm (a join).  Prints us per link."""
import sys, time, torch
dev = torch.device("cuda")
a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
c = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
tiny = torch.zeros(1024, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
N = 200


def run(mode):
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    h0 = time.perf_counter()
    pend = None
    for _ in range(N):
        if mode == "join" and pend is not None:
            main.wait_event(pend)
        torch.mm(a, b, out=c)
        if mode != "plain":
            ev = torch.cuda.Event(); ev.record(main)
            if mode in ("wait", "join"):
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    tiny.add_(1.0)
                    if mode == "join":
                        pend = torch.cuda.Event(); pend.record(side)
    h1 = time.perf_counter()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / N, (h1 - h0) * 1e6 / N


for m in ("plain", "record", "wait", "join", "plain"):
    run(m)
    g, h = run(m)
    print(f"{m:7s}: {g:8.1f} us per link on the GPU, {h:6.1f} us of host time per link")
