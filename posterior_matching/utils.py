"""reference import path posterior_matching.utils"""
from posterior_matching_amd.utils import *  # noqa: F401,F403
from posterior_matching_amd.utils import (configure_environment, load_datasets, cyclical_annealing_schedule,  # noqa: F401
                                          make_run_dir, TensorBoardCallback)
