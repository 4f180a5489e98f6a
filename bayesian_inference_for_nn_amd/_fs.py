"""Small filesystem helper shared by Optimizer.train (model_save_path) and BayesianModel.store: both start
from an emptied target directory, as the reference does (Optimizer.py:78-88, BayesianModel.py:166-176)."""

import os
import shutil


def empty_folder(path: str) -> None:
    """Removes everything inside `path` (files, links, sub-directories); a failure is reported, not raised."""
    with os.scandir(path) as entries:
        for entry in entries:
            try:
                if entry.is_dir(follow_symlinks=False):
                    shutil.rmtree(entry.path)
                else:
                    os.unlink(entry.path)
            except OSError as e:
                print(f"Failed to delete {entry.path}. Reason: {e}")
