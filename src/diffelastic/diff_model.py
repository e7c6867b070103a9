from diffsound_amd.diffelastic.diff_model import *  # noqa: F401,F403
from diffsound_amd.diffelastic.diff_model import (DiffSoundObj, FixedLinear, Material, MatSet, TetMesh,  # noqa: F401
                                                  TrainableLinear, build_model)
