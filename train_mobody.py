"""Repo-root entry point with the reference's script name: `python train_mobody.py --policy MOBODY --env walker2d-friction ...`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mobody_amd.train_mobody import main  # noqa: E402

if __name__ == "__main__":
    main()
