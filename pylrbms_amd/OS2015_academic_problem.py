"""OS2015 academic problem (reference python/dune/pylrbms/OS2015_academic_problem.py:19-67):
Omega = [-1,1]^2, lambda(mu) = 1 + (1 - mu) cos(pi/2 x) cos(pi/2 y), kappa = I, f = pi^2/2 cos cos."""
from pylrbms_amd.functions import make_constant_function_2x2, make_expression_function_1x1
from pylrbms_amd.grid import grid_info, make_boundary_info, make_grid
from pylrbms_amd.parameters import ExpressionParameterFunctional


def init_grid_and_problem(config, mu_bar=1, mu_hat=1, mpi_comm=None):
    lower_left, upper_right = [-1, -1], [1, 1]
    inner_boundary_id = 18446744073709551573
    grid = make_grid((lower_left, upper_right), config['num_subdomains'],
                     config['half_num_fine_elements_per_subdomain_and_dim'], inner_boundary_id, mpi_comm=mpi_comm)
    all_dirichlet_boundary_info = make_boundary_info(grid, {'type': 'xt.grid.boundaryinfo.alldirichlet'})
    cos = '(cos(0.5*pi*x[0])*cos(0.5*pi*x[1]))'
    diffusion_functions = [make_expression_function_1x1(grid, 'x', '1+{}'.format(cos), order=2, name='lambda_0'),
                           make_expression_function_1x1(grid, 'x', '-1*{}'.format(cos), order=2, name='lambda_1')]
    parameter_type = {'diffusion': (1,)}
    coefficients = [ExpressionParameterFunctional('1.', parameter_type),
                    ExpressionParameterFunctional('diffusion', parameter_type)]
    kappa = make_constant_function_2x2(grid, [[1., 0.], [0., 1.]], name='kappa')
    f = make_expression_function_1x1(grid, 'x', '0.5*pi*pi*{}'.format(cos), order=2, name='f')
    mbc = '1+(1-{})*{}'.format(mu_bar, cos)
    lambda_bar = make_expression_function_1x1(grid, 'x', mbc, order=2, name='lambda_bar')
    # as written in the reference (OS2015_academic_problem.py:48-50): lambda_hat is built from the mu_bar expression too
    lambda_hat = make_expression_function_1x1(grid, 'x', mbc, order=2, name='lambda_hat')
    return {'grid': grid,
            'mpi_comm': mpi_comm,
            'boundary_info': all_dirichlet_boundary_info,
            'inner_boundary_id': inner_boundary_id,
            'lambda': {'functions': diffusion_functions, 'coefficients': coefficients},
            'lambda_bar': lambda_bar,
            'lambda_hat': lambda_hat,
            'kappa': kappa,
            'f': f,
            'parameter_type': parameter_type,
            'mu_bar': (mu_bar,),
            'mu_hat': (mu_hat,),
            'mu_min': (min(0.1, mu_bar, mu_hat),),
            'mu_max': (max(1, mu_bar, mu_hat),),
            'parameter_range': (min(0.1, mu_bar, mu_hat), max(1, mu_bar, mu_hat))}
