"""Helpers shared by bidders and the coordinator."""


def convert_marginal_costs_to_actual_costs(power_marginal_cost_pairs):
    """[(MW, $/MWh)...] sorted by power -> [(MW, $)...]: cumulative marginal cost x delta power.

    Restates idaes.apps.grid_integration.utils.convert_marginal_costs_to_actual_costs, called by the reference
    at dispatches/workflow/coordinator.py:65 and renewables_case/PEM_parametrized_bidder.py:65,103; pinned by
    test_wind_PEM_double_loop.py:211-213 (33.72 = 1.1238 MW x 30 $/MWh).
    """
    out = []
    cost = prev = 0.0
    for k, (power, marginal) in enumerate(power_marginal_cost_pairs):
        cost = power * marginal if k == 0 else cost + (power - prev) * marginal
        out.append((power, cost))
        prev = power
    return out
