"""Drop-in `models` package: same entry points as the reference's models/__init__.py:25-67
(find_model_using_name / get_option_setter / create_model) over the HIP-backed model classes."""
import importlib

from .base_model import BaseModel


def find_model_using_name(model_name):
    modellib = importlib.import_module('.' + model_name + '_model', package=__name__)
    target = model_name.replace('_', '') + 'model'
    model = None
    for name, cls in modellib.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type) and issubclass(cls, BaseModel):
            model = cls
    if model is None:
        print("In %s_model.py, there should be a subclass of BaseModel with class name that matches %s in lowercase." % (model_name, target))
        exit(0)
    return model


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
    instance = find_model_using_name(opt.model)(opt)
    print("model [%s] was created" % type(instance).__name__)
    return instance
