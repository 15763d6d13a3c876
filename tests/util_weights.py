from hyperseg_amd.utils.synthetic import fill_by_name, tensor_for  # noqa: F401
