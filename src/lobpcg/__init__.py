from diffsound_amd.lobpcg import lobpcg, lobpcg_func  # noqa: F401
