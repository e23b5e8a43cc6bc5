import upsnet_amd.utils   # noqa: F401  (submodules come from the alias finder installed by lib/__init__.py)

__path__ = []
