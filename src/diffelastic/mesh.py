from diffsound_amd.diffelastic.mesh import TetMesh, read_gmsh22, write_gmsh22  # noqa: F401
from diffsound_amd.diffelastic.mesh import largest_connected_component  # noqa: F401,E402
