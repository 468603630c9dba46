"""Pruner registry — the reference's registration surface for this path
(LAVIS/lavis/common/registry.py:113-137 `register_pruner`, :270-271 `get_pruner_class`)."""


class Registry:
    mapping = {"pruner_name_mapping": {}}

    @classmethod
    def register_pruner(cls, name):
        def wrap(pruner_cls):
            from .pruners.base_pruner import BasePruner
            assert issubclass(pruner_cls, BasePruner), "All pruners must inherit BasePruner class"
            if name in cls.mapping["pruner_name_mapping"]:
                raise KeyError("Name '{}' already registered for {}.".format(
                    name, cls.mapping["pruner_name_mapping"][name]))
            cls.mapping["pruner_name_mapping"][name] = pruner_cls
            return pruner_cls
        return wrap

    @classmethod
    def get_pruner_class(cls, name):
        return cls.mapping["pruner_name_mapping"].get(name, None)

    @classmethod
    def list_pruners(cls):
        return sorted(cls.mapping["pruner_name_mapping"].keys())


registry = Registry()
