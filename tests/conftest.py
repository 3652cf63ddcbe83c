import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (authoring container only)')


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir('/root/reference/nlf')
    skip_ref = pytest.mark.skip(reason='/root/reference is not present on this machine')
    for item in items:
        if 'reference' in item.keywords and not have_ref:
            item.add_marker(skip_ref)
