"""Nested-class configs (mirror of reference legged_gym/envs/base/base_config.py:33-55).

Instantiating a config turns every nested *class* attribute into an *instance* of that class,
recursively, so `cfg.env.num_envs` can be mutated per object (task_registry relies on that).
"""
import inspect


class BaseConfig:
    def __init__(self) -> None:
        self.init_member_classes(self)

    @staticmethod
    def init_member_classes(obj):
        for key in dir(obj):
            if key == "__class__":
                continue
            member = getattr(obj, key)
            if inspect.isclass(member):
                inst = member()
                setattr(obj, key, inst)
                BaseConfig.init_member_classes(inst)
