"""Fast3DGenerator: the class the add-on takes from `StableFast/generate.py` -- same constructor argument, public
attributes (`checkpoint_dir`, `texture_resolution`, `image_path`, `device`, `model`), keyword arguments and return
codes (generate.py:8-59), running on the MI355X kernels."""
import os

import torch

from .._facade import STATUS_NOT_LOADED, STATUS_OK, GeneratorFacade
from .system import SF3D

ROOT_DIR = os.path.dirname(os.path.abspath(__file__))


class Fast3DGenerator(GeneratorFacade):
    def __init__(self, device):
        super().__init__(device, checkpoint_dir=ROOT_DIR + "/checkpoints/", texture_resolution=1024)
        self.last_mesh = None  # headless callers read the result dict here

    def _construct_model(self):
        model = SF3D.from_pretrained(self.checkpoint_dir, config_name="config.yaml", weight_name="model.safetensors",
                                     device=self.device)
        return model.to(self.device).eval()

    def generate_mesh(self, input_image, input_name=None, remesh_option="triangle", texture_resolution=512,
                      vertex_simplification_factor="high", enable_texture=True):
        """Exceptions propagate: the reference's try/except around this body is commented out (generate.py:38,57-59)."""
        if self.model is None:
            return STATUS_NOT_LOADED
        torch.cuda.empty_cache()
        if remesh_option != "none" and self.model.remesher is None:
            # the add-on always asks for 'triangle' (generate.py:33); with the remesher switched off hand over the
            # un-remeshed mesh
            print("[Generation Warning] no remesher set: remesh_option=%r ignored, the mesh keeps its "
                  "marching-tetrahedra resolution" % remesh_option)
            remesh_option = "none"
        result, _global = self.model.run_image(input_image, bake_resolution=texture_resolution, remesh=remesh_option,
                                               vertex_simplification_factor=vertex_simplification_factor,
                                               enable_texture=enable_texture)
        if result is None:
            raise Exception("Mesh shape was zero")
        self.last_mesh = result
        if _blender_available():
            self.model.import_mesh_blender(result, input_name)
        return STATUS_OK


def _blender_available():
    try:
        import bpy  # noqa: F401
    except ImportError:
        return False
    return True
