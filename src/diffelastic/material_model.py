from diffsound_amd.diffelastic.material_model import Material, MatSet  # noqa: F401
