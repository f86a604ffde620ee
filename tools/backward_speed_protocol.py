"""The reference's own backward timing protocol (test/backward_speed.py:22-69: model + a 1x1 convolution head, loss = head(output).sum(),
SGD over both; 5 warm-up steps, 10 timed steps, device events around forward + backward + optimizer step, a synchronize and
torch.cuda.empty_cache() before every step) applied to this implementation.  Like-for-like with BASELINE.md's 163.08 ms / 6016.5 MB
(image 448^2, 384 x 28^2 features -> 448^2, default NAF() with window 9, fp32 tensors as in test/test_utils.py:78-82).
Arms: fp32 as the reference's harness runs it; under torch.autocast(bfloat16), the reference's own use_bf16 training mode
(train.py:120), where the stem runs on the library's HIP kernels; each with the reference's protocol and back to back.
--profile: torch.profiler table of three autocast steps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import naf_amd

dev = torch.device("cuda:0")
NUM_RUNS = 10
POINTS = {"REF448 (reference's point)": (384, 28, 448, 9),
          "P14 (448^2, patch-14 grid)": (384, 32, 448, 9)}      # a DINOv2-S/14 backbone's 32 x 32 tokens: 14-pixel cells (round 6: cell backward with partial row tiles)
if "--all" in sys.argv:
    POINTS.update({"448^2 C768": (768, 28, 448, 9), "448^2 C1024 k7": (1024, 28, 448, 7), "G1 (1024^2, C768, k7)": (768, 64, 1024, 7)})
for name, (C, lr, out, ks) in POINTS.items():
    for arm in ("fp32", "autocast bf16"):
        torch.manual_seed(0)
        model = naf_amd.NAF(kernel_size=ks).to(dev)          # train mode, as ModelWrapper leaves it
        head = torch.nn.Conv2d(C, 1, 1).to(dev)
        opt = torch.optim.SGD(list(model.parameters()) + list(head.parameters()), lr=0.01)
        img = torch.randn(1, 3, out, out, device=dev)
        ft = torch.randn(1, C, lr, lr, device=dev)

        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(arm != "fp32")):
                o = model(img, ft, (out, out))
                loss = head(o).sum()
            loss.backward()
            opt.step()
            return loss

        for variant in ("reference protocol (empty_cache + sync per step)", "back to back"):
            ref = variant.startswith("reference")
            for _ in range(5):
                if ref: torch.cuda.empty_cache()
                step()
                if ref: torch.cuda.empty_cache()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
            if ref:
                total = 0.0
                for _ in range(NUM_RUNS):
                    torch.cuda.empty_cache(); torch.cuda.synchronize()
                    s.record(); step(); e.record(); torch.cuda.synchronize()
                    torch.cuda.empty_cache()
                    total += s.elapsed_time(e)
            else:
                s.record()
                for _ in range(NUM_RUNS): step()
                e.record(); torch.cuda.synchronize()
                total = s.elapsed_time(e)
            print("%-28s %-14s %-50s %8.3f ms per step   peak %6.0f MB" % (name, arm, variant, total / NUM_RUNS, torch.cuda.max_memory_allocated() / 2**20), flush=True)
        if "--profile" in sys.argv and arm != "fp32":
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
                for _ in range(3): step()
                torch.cuda.synchronize()
            print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=80))
print("(reference, A100-40GB, its protocol, fp32: 163.08 ms, 6016.5 MB at REF448 -- test/test_results.json:250-251)")
