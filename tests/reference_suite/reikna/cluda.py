def cuda_id():
    return 'cuda'


def ocl_id():
    return 'ocl'
