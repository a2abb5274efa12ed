"""name -> class registries selected by the `type:` strings of the yaml configs.

Same registry set as the reference (common/utils/registry.py:31-86); only the ones the hot path uses are populated.
"""


class Registry:
    def __init__(self, name):
        self._name = name
        self._items = {}

    def register(self, obj=None):
        def add(o):
            key = o.__name__
            if key in self._items:
                raise KeyError("'{}' is already registered in registry '{}'".format(key, self._name))
            self._items[key] = o
            return o

        return add if obj is None else add(obj)

    def get(self, name):
        if name not in self._items:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return self._items[name]

    def __contains__(self, name):
        return name in self._items

    def keys(self):
        return self._items.keys()


MODEL_REGISTRY = Registry('model')
ENCODER_REGISTRY = Registry('encoder')
MODULE_REGISTRY = Registry('module')
BOUND_REGISTRY = Registry('bound')
LOSS_REGISTRY = Registry('loss')
METRIC_REGISTRY = Registry('metric')
DATASET_REGISTRY = Registry('dataset')
