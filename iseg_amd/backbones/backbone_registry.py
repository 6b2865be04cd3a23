"""backbones/backbone_registry.py of the reference (:2-18): name(s) -> constructor; default name is the lower-cased
class name; the first registration of a name wins."""
backbone_registry_dict = {}


def register_backbone(backbone_class, name=None):
    if name is None:
        name = backbone_class.__name__
        name = name.lower()
    if isinstance(name, tuple):
        name = list(name)
    if not isinstance(name, list):
        name = [name]
    for n in name:
        if n not in backbone_registry_dict:
            backbone_registry_dict[n] = backbone_class
