"""TripoGenerator: what `GUIPanel.py:11` imports from `TripoSR/generate.py` -- same constructor argument, public
attributes (`checkpoint_dir`, `chunk_size`, `mc_resolution`, `image_path`, `device`, `model`) and return codes
(generate.py:9-43), running on the MI355X kernels."""
import os

import torch

from ._facade import STATUS_FAILED, STATUS_NOT_LOADED, STATUS_OK, GeneratorFacade
from .tsr.system import TSR

ROOT_DIR = os.path.dirname(os.path.abspath(__file__))


class TripoGenerator(GeneratorFacade):
    def __init__(self, device):
        super().__init__(device, checkpoint_dir=ROOT_DIR + "/checkpoints/", chunk_size=8192, mc_resolution=256)
        self.last_meshes = None  # headless callers read the result here (inside Blender it goes to the scene)

    def _construct_model(self):
        model = TSR.from_pretrained(self.checkpoint_dir, config_name="config.yaml", weight_name="model.ckpt")
        model.renderer.set_chunk_size(self.chunk_size)
        return model.to(self.device)

    def generate_mesh(self, input_image, input_name=None, enable_texture=False):
        if self.model is None:
            return STATUS_NOT_LOADED
        try:
            with torch.no_grad():
                codes = self.model([input_image], device=self.device)
            self.last_meshes = self.model.extract_mesh(codes, enable_texture=enable_texture, mesh_name=input_name,
                                                       resolution=self.mc_resolution)
        except Exception as err:
            print(self.run_error_tag, err)
            return STATUS_FAILED
        return STATUS_OK
