"""Shim for setuptools < 61 and `python setup.py egg_info`; all metadata lives in pyproject.toml / setup.cfg."""
from setuptools import setup

setup()
