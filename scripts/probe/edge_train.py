import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_train as T
bad = 0
for (V, nt, B, tied) in [(40, 30, 5, False), (33, 20, 256, False), (100, 64, 1, True), (257, 200, 33, False), (64, 64 - 8, 64, True), (31, 20, 3, False)]:
    try:
        T.test_train_step_gradients(V, nt, 256, B, tied, 0.0, 0.75, 0.8)
        try:
            T.test_train_step_bf16_gemms(V, nt, 256, B, tied)
        except AssertionError as e:
            print("bf16 vs f32 check", V, nt, B, tied, "assert", repr(e))
        print("ok", V, nt, B, tied, flush=True)
    except Exception as e:
        bad += 1; print("FAILED", V, nt, B, tied, repr(e)[:300], flush=True)
print("edge cases bad:", bad)
