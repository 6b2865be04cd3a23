from .intern_image import InternImage, intern_image_base, intern_image_small, intern_image_tiny  # noqa: F401
