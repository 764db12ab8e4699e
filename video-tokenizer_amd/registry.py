"""Name -> class registry with the reference's semantics (/root/reference/models/models.py:5-27):
`register(name)` decorator, `make(spec, args=None, load_sd=False)` which merges `args` over a deep copy
of spec['args'], drops kwargs the class does not accept unless it declares **kwargs, and optionally
loads spec['sd'] strictly."""
import copy
import inspect

models = {}


def register(name):
    def decorator(cls):
        models[name] = cls
        return cls
    return decorator


def make(model_spec, args=None, load_sd=False):
    if args is not None:
        model_args = copy.deepcopy(model_spec["args"])
        model_args.update(args)
    else:
        model_args = model_spec["args"]
    cls = models[model_spec["name"]]
    params = inspect.signature(cls).parameters
    if "kwargs" not in params:
        model_args = {k: v for k, v in model_args.items() if k in params}
    model = cls(**model_args)
    if load_sd:
        model.load_state_dict(model_spec["sd"], strict=True)
    return model
