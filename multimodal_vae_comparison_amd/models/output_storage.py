"""VAEOutput / ModalityOutput: the typed container `forward()` returns (reference: models/output_storage.py).
Same field names and `unpack_values()` order; the per-set isinstance scan over every torch.distributions class
(output_storage.py:48-52) is replaced by one isinstance against the Distribution base class."""
import torch

fields = ["encoder_dist", "joint_dist", "joint_decoder_dist", "decoder_dist", "dec_dist_private", "latent_samples",
          "enc_dist_private", "cross_decoder_dist"]


class ModalityOutput:
    def __init__(self, id: str):
        self.id = id
        for f in fields:
            setattr(self, f, None)

    def set_value(self, field: str, val):
        if val is not None:
            self.check_field_valid(field)
            if field not in ["latent_samples", "cross_decoder_dist"]:
                self.check_is_distribution(val, field)
            else:
                assert isinstance(val, dict), "Expected {} to be a dict! Got {}".format(field, val)
        setattr(self, field, val)

    def get_value(self, field: str):
        self.check_field_valid(field)
        return getattr(self, field)

    @staticmethod
    def check_is_distribution(val, field):
        assert isinstance(val, torch.distributions.Distribution), \
            "{} value must be an instance of torch.distributions! Got: {}".format(field, val)

    @staticmethod
    def check_field_valid(field: str):
        assert field in fields, "Unsupported field name {}. Choose out of: {}".format(field, fields)


class VAEOutput:
    def __init__(self):
        self.mods = {}

    def add_new_modality(self, name: str):
        self.mods[name] = ModalityOutput(name)

    def set_value(self, mod: str, field: str, val):
        if mod not in self.mods:
            self.add_new_modality(mod)
        self.mods[mod].set_value(field, val)

    def set_with_dict(self, d: dict, field: str):
        if d is not None:
            for key in d.keys():
                self.set_value(key, field, d[key])

    def set_to_all(self, field: str, val):
        for key in self.mods.keys():
            self.set_value(key, field, val)

    def get_all_values(self, field):
        return [m.get_value(field) for m in self.mods.values()]

    def unpack_values(self):
        return {f: self.get_all_values(f) for f in fields}
