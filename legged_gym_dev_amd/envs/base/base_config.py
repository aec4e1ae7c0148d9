"""Nested-class configuration objects.

Mirrors the contract of the reference's BaseConfig (legged_gym/envs/base/base_config.py:38-55):
constructing a config turns every member *class* (recursively) into an *instance*, so
``cfg.env.num_envs`` works on instances and per-task configs override by subclassing the
nested classes.
"""
import inspect


def _instantiate_members(node):
    for attr in dir(node):
        if attr == "__class__":
            continue
        member = getattr(node, attr)
        if inspect.isclass(member):
            inst = member()
            setattr(node, attr, inst)
            _instantiate_members(inst)


class BaseConfig:
    def __init__(self) -> None:
        _instantiate_members(self)

    # kept for API compatibility with code that calls BaseConfig.init_member_classes(obj)
    init_member_classes = staticmethod(_instantiate_members)


class S:
    """One configuration section in a :func:`cfg_class` tree: keyword arguments become class attributes,
    nested ``S`` values become nested sections."""

    def __init__(self, **attrs):
        self.attrs = attrs


def _section(name, parent_section, spec, qual):
    bases = (parent_section,) if parent_section is not None else ()
    body = {"__qualname__": qual, "__module__": spec.attrs.get("__module__", __name__)}
    for key, value in spec.attrs.items():
        if isinstance(value, S):
            body[key] = _section(key, getattr(parent_section, key, None) if parent_section is not None else None, value,
                                 f"{qual}.{key}")
        else:
            body[key] = value
    return type(name, bases, body)


def cfg_class(name, base, tree, doc=None, module=None):
    """Build the configuration class ``name(base)`` from a declarative tree.

    ``tree`` maps section names to :class:`S` specs (or plain attributes to values).  Every section becomes a
    nested class deriving from the section of the same name in ``base`` (when there is one), exactly what a
    hand-written ``class env(Base.env): ...`` does -- so tasks keep overriding by subclassing, and
    ``BaseConfig.__init__`` instantiates the sections as before."""
    body = {"__doc__": doc, "__module__": module or __name__}
    for key, value in tree.items():
        if isinstance(value, S):
            body[key] = _section(key, getattr(base, key, None), value, f"{name}.{key}")
        else:
            body[key] = value
    return type(name, (base,), body)
