from .lib import load_library, library_path, HipKernelError  # noqa: F401
