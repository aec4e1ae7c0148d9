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
