"""Shared behaviour of the two generator facades the add-on instantiates (TripoSR/generate.py::TripoGenerator and
StableFast/generate.py::Fast3DGenerator): lazy model construction with the reference's integer status codes."""

STATUS_OK = 0
STATUS_NOT_LOADED = 1
STATUS_FAILED = 2


class GeneratorFacade:
    """Sub-classes provide `_construct_model()` (returns the model, already on `self.device`).

    Status codes (the add-on branches on them, GUIPanel.py:195-220):
      initiate_model  -> 0 loaded now | 2 construction failed (message printed) | None: a model was already there
      generate_mesh   -> 0 done       | 1 no model yet        | 2 generation failed (message printed)
    """

    init_error_tag = "[Model Dos Initialization Error]"
    run_error_tag = "[Generation Error]"

    def __init__(self, device, **attributes):
        self.device = device
        self.model = None
        self.image_path = ""
        for key, value in attributes.items():
            setattr(self, key, value)

    def _construct_model(self):  # pragma: no cover (abstract)
        raise NotImplementedError

    def initiate_model(self):
        if self.model is not None:
            return None
        try:
            self.model = self._construct_model()
        except Exception as err:
            self.model = None
            print(self.init_error_tag, err)
            return STATUS_FAILED
        return STATUS_OK
