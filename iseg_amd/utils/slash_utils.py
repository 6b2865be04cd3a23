"""utils/slash_utils.py of the reference: Keras 3 refuses '/' in layer names, so stored names carry '.' instead."""
REPLACE_SLASH = True


def replace_slash(name):
    if REPLACE_SLASH and name is not None:
        name = name.replace("/", ".")
    return name
