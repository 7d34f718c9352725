from .system import TSR, Mesh, MarchingCubeHelper, TriplaneNeRFRenderer, param_spec, load_config, DEFAULT_CFG  # noqa: F401
