"""A/B timing of alternative builds of the library on one box: each variant runs in a child process, interleaved twice.
usage: gpu_ab.py libA.so libB.so ..."""
import subprocess, sys, os
here = os.path.dirname(os.path.abspath(__file__))
libs = sys.argv[1:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ, TGR_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, os.path.join(here, "gpu_quickbench.py"), "4096", "tensegrity_env-v0", "150", "1"], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("n=")]
        print(os.path.basename(lib), line[-1][:230] if line else out.stderr[-300:], flush=True)
