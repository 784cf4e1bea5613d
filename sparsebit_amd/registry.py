"""A keyed class registry shared by quantizers, observers and sparsers.

The reference keeps three module-level dicts filled by three decorator functions
(quantizers/__init__.py:1-6, observers/__init__.py:1-6, sparse/sparsers/__init__.py:1-8);
plugins rely on two properties of them, which `Registry` keeps: the key is the lower-cased
class attribute, and registering a key again replaces the earlier class.
"""


class Registry(dict):
    def __init__(self, what, key_attr):
        super().__init__()
        self.what = what
        self.key_attr = key_attr

    def register(self, cls):
        self[getattr(cls, self.key_attr).lower()] = cls
        return cls

    def resolve(self, name):
        try:
            return self[name.lower()]
        except KeyError:
            raise AssertionError("no found an implement of {} (known {}s: {})".format(name, self.what, sorted(self)))


def impl_type(obj):
    """The sparsebit_amd class behind an object.  plugin.install() registers classes derived from
    (sparsebit_amd class, reference base) into the reference so that its isinstance checks hold; exact-type
    dispatch ("plain LSQ, not a subclass that transforms its inputs") must look through that derivation."""
    t = type(obj)
    return t.__dict__.get("_sbq_impl", t)
