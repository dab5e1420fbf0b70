import sys, os, runpy, cProfile, pstats, torch
torch.autograd.set_multithreading_enabled(False)
sys.argv = ["train_step.py", "--steps", "10", "--warmup", "2", "--small"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts", "train_step.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(38)
