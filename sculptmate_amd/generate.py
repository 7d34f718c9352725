"""TripoGenerator: what `GUIPanel.py:11` imports from `TripoSR/generate.py` -- same constructor argument, public
attributes (`checkpoint_dir`, `chunk_size`, `mc_resolution`, `image_path`, `device`, `model`) and return codes
(generate.py:9-43), running on the MI355X kernels."""
import os

import torch

from ._facade import STATUS_FAILED, STATUS_NOT_LOADED, STATUS_OK, GeneratorFacade
from .tsr.system import TSR

ROOT_DIR = os.path.dirname(os.path.abspath(__file__))


class TripoGenerator(GeneratorFacade):
    """One attribute beyond the reference's: `precision`, the arithmetic of the transformer, read when the model is constructed
    (initiate_model):
      "bf16"   (default; BASELINE config 2) bf16 storage, fp32 accumulate -- the fast mode; the scene code moves by ~0.8 % against
               the fp32 reference, i.e. the mesh is within 1e-4 of it only GIVEN the same scene code;
      "fp16l2" the faster of the two modes that meet the 1e-4 vertex tolerance against the reference's fp32 CPU path end to end:
               fp32 storage, the Linears and the backbone's attention with two fp16 limbs per operand (22 bits), fp32 accumulate
               (~2.4x the forward time); an image whose activations leave the fp16 range is redone on three bf16 limbs;
      "bf16l3" the same tolerance with the fp32 exponent range: every matrix product with both operands split exactly into three
               bf16 limbs, fp32 accumulate (~3.6x the forward time);
      "fp32"   the exact-fp32 matrix instruction (slowest; the parity yard-stick).
    The environment variable SCULPT_PRECISION overrides the default for an add-on that cannot be edited."""

    def __init__(self, device):
        super().__init__(device, checkpoint_dir=ROOT_DIR + "/checkpoints/", chunk_size=8192, mc_resolution=256,
                         precision=os.environ.get("SCULPT_PRECISION", "bf16"))
        self.last_meshes = None  # headless callers read the result here (inside Blender it goes to the scene)

    def _construct_model(self):
        model = TSR.from_pretrained(self.checkpoint_dir, config_name="config.yaml", weight_name="model.ckpt",
                                    precision=self.precision)
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
