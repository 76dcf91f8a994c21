import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R + "/sp-gan_amd", R + "/tests", R]
import pytest
from spgan import ops
os.chdir(R)
with ops.nt_tile_hint(int(sys.argv[1])):
    sys.exit(pytest.main(["tests/test_parity_gpu.py", "-q", "-m", "gpu", "-k", "test_train_step_golden", "-x", "--no-header", "-p", "no:cacheprovider"]))
