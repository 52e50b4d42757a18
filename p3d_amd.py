"""Importable alias of the product package (its directory name is not a valid Python identifier)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module('3d-pose-estimation-with-previleged-information_amd')
