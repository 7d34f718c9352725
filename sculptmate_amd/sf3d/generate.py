"""Fast3DGenerator -- drop-in for /root/reference/StableFast/generate.py (same constructor, attributes, methods and
integer return codes)."""
import os

import torch

from .system import SF3D

ROOT_DIR = os.path.dirname(os.path.abspath(__file__))


class Fast3DGenerator():
    def __init__(self, device):
        self.checkpoint_dir = ROOT_DIR + '/checkpoints/'
        self.texture_resolution = 1024
        self.image_path = ''
        self.device = device
        self.model = None
        self.last_mesh = None

    def initiate_model(self):
        """0 ok | 2 error | None if already loaded (generate.py:16-30)."""
        if self.model is None:
            try:
                self.model = SF3D.from_pretrained(
                    self.checkpoint_dir,
                    config_name="config.yaml",
                    weight_name="model.safetensors",
                    device=self.device
                )
                self.model.to(self.device)
                self.model.eval()
            except Exception as e:
                self.model = None
                print('[Model Dos Initialization Error]', e)
                return 2
            return 0

    def generate_mesh(self, input_image, input_name=None,
                      remesh_option='triangle',
                      texture_resolution=512,
                      vertex_simplification_factor='high',
                      enable_texture=True):
        """0 ok | 1 model not loaded; exceptions propagate (the reference's try/except is commented out,
        generate.py:38,57-59)."""
        if self.model is None:
            return 1
        torch.cuda.empty_cache()
        mesh, glob_dict = self.model.run_image(
            input_image,
            bake_resolution=texture_resolution,
            remesh=remesh_option,
            vertex_simplification_factor=vertex_simplification_factor,
            enable_texture=enable_texture
        )
        if mesh is None:
            raise Exception('Mesh shape was zero')
        self.last_mesh = mesh
        try:
            import bpy  # noqa: F401
        except ImportError:
            return 0  # headless: the mesh dict stays in self.last_mesh
        self.model.import_mesh_blender(mesh, input_name)
        return 0
