"""Drop-in alias: ``import u2pl...`` resolves to the MI355X-native ``u2pl_amd``
package so the reference's YAML dotted paths (``u2pl.models.resnet.resnet101``)
and imports (``from u2pl.utils.loss_helper import ...``) keep working."""
import importlib
import sys

import u2pl_amd

for _name in ["models", "models.base", "models.resnet", "models.decoder", "models.model_helper", "utils",
              "utils.loss_helper", "utils.utils", "utils.lr_helper", "utils.dist_helper", "dataset", "dataset.builder",
              "dataset.augmentation"]:
    try:
        sys.modules["u2pl." + _name] = importlib.import_module("u2pl_amd." + _name)
    except ImportError:
        pass
__path__ = u2pl_amd.__path__
